"""Developer check: circuit time of the n-qubit depth-40 RQC vs the fusion width (compress)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import rqc_1q2q  # noqa: E402
from hybridq_amd.simulation import EvolutionState, _execute_ops, _plan_ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ct = sys.argv[2] if len(sys.argv) > 2 else 'complex64'
core.set_stream(torch.cuda.current_stream().cuda_stream)
gates = rqc_1q2q(n, depth=40, seed=n)
st = EvolutionState(list(range(n)), complex_type=ct, initial_state='0' * n)
for compress in (0, 2, 3, 4, 5, 6):
    t0 = time.perf_counter()
    ops = _plan_ops(gates, st.qubits, n, st.complex_type, compress, False)
    t_plan = time.perf_counter() - t0
    _execute_ops(st, ops)
    core.sync()
    t0 = time.perf_counter()
    _execute_ops(st, ops)
    core.sync()
    el = time.perf_counter() - t0
    hist = {}
    for qs, U in ops:
        hist[len(qs)] = hist.get(len(qs), 0) + 1
    print(f'n={n} {ct} compress={compress}: {len(ops):4d} calls {dict(sorted(hist.items()))}  {1e3 * el:8.1f} ms/circuit  '
          f'{len(gates) / el:8.1f} logical gate-apps/s  (fusion {t_plan:.2f} s)', flush=True)
