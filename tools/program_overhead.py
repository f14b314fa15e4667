"""Developer check: eager per-gate calls vs a compiled Program (loop replay / hipGraph replay)
on the RQC at small n, where one gate kernel is shorter than a Python-level call."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.circuits import rqc_1q2q  # noqa: E402
from hybridq_amd.simulation import EvolutionState, _execute_ops, _plan_ops  # noqa: E402

core.set_stream(torch.cuda.current_stream().cuda_stream)
for n in (12, 16, 20, 22, 24):
    gates = rqc_1q2q(n, depth=20, seed=1)
    for label, kw in (('per-gate', dict(compress=0)), ('compress=4', dict(compress=4)), ('blocked', dict(blocked=True))):
        st = EvolutionState(list(range(n)), complex_type='complex64', initial_state='0' * n)
        ops = _plan_ops(gates, st.qubits, n, st.complex_type, kw.get('compress', 4), kw.get('blocked', False))
        reps = 10
        _execute_ops(st, ops)
        core.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            _execute_ops(st, ops)
        core.sync()
        t_eager = (time.perf_counter() - t0) / reps
        res = {}
        for graph in ('0', '1'):
            os.environ['HQ_PROGRAM_GRAPH'] = graph
            prog = st.compile(gates, **kw)
            prog.run()
            prog.run()
            core.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                prog.run()
            core.sync()
            res[graph] = (time.perf_counter() - t0) / reps
            prog.free()
        print(f'n={n:2d} {label:10s} ops={len(ops):4d}  eager {1e3 * t_eager:8.3f} ms   program(loop) {1e3 * res["0"]:8.3f} ms'
              f'   program(graph) {1e3 * res["1"]:8.3f} ms   speed-up {t_eager / res["1"]:5.2f}x', flush=True)
