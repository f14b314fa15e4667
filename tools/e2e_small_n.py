"""End-to-end wall time of simulate() against its own loop time for the state sizes the reference is typically run at
(n = 16..26, BASELINE config-2 generator at depth 40): how much of the call is host-side schedule planning."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd.circuits import rqc_1q2q  # noqa: E402
from hybridq_amd.simulation import simulate  # noqa: E402

for n in (16, 20, 24, 26, 28):
    gates = rqc_1q2q(n, depth=40, seed=n)
    for opt, kw in (('evolution', {}), ('evolution-hybridq', {}), ('evolution', dict(compress=0)), ('evolution', dict(blocked=True))):
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            psi, info = simulate(gates, initial_state='0' * n, optimize=opt, return_info=True, qubits=list(range(n)),
                                 simplify=False, return_numpy_array=False, **kw)
            wall = time.perf_counter() - t0
            if best is None or wall < best[0]:
                best = (wall, info)
        wall, info = best
        sch = info.get('schedule', {})
        print(f"n={n} {opt:18s} {str(kw):18s} wall {wall * 1e3:8.1f} ms  loop {info['runtime (s)'] * 1e3:8.2f} ms  passes {info['n_passes']:4d}  "
              f"{sch.get('chosen', '')} not_planned={sch.get('not_planned', '')}", flush=True)
