"""BASELINE config 5 on one GPU: a 15-qubit noisy circuit = 30-qubit state vector through
hybridq_amd.dm (unitaries on both copies + depolarizing superoperators), fused vs blocked."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd.circuits import rqc_1q2q  # noqa: E402
from hybridq_amd.dm import depolarizing, simulate  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 15
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 10
circ = []
for U, qs in rqc_1q2q(nq, depth=depth, seed=nq):
    circ.append((U, qs))
    circ.append(depolarizing(qs, 0.01 if len(qs) == 1 else 0.02))
out = {'n_qubits': nq, 'statevector_qubits': 2 * nq, 'items': len(circ)}
for name, kw in (('fused_k4', dict(compress=4)), ('blocked', dict(blocked=True))):
    simulate(circ, initial_state='0', complex_type='complex64', return_numpy_array=False, **kw)  # warm-up
    st, info = simulate(circ, initial_state='0', complex_type='complex64', return_numpy_array=False,
                        return_info=True, **kw)
    # trace(rho) = sum_x rho[x, x]: left index = high half; sample it on the device
    import torch
    idx = torch.arange(1 << nq, device='cuda') * ((1 << nq) + 1)
    tr = float(st.planes[0][idx].double().sum())
    out[name] = {'runtime_s': info['runtime (s)'], 'passes': info['n_passes'], 'trace_rho': tr,
                 'amplitude_updates_per_s': info['n_passes'] * float(1 << (2 * nq)) / info['runtime (s)']}
print(json.dumps(out))
