"""Host-side planning time of simulate() for the benchmark circuit (no GPU needed): simplify, the fused schedules, the
cache-blocked planner (quick and full search), and what choose_schedule as a whole costs.
    python tools/plan_time.py [n] [depth]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import simulation as sim  # noqa: E402
from hybridq_amd.blocking import blocked_stats  # noqa: E402
from hybridq_amd.circuits import rqc_1q2q  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 40
gates = rqc_1q2q(n, depth=depth, seed=1)
qubits = list(range(n))
ct = np.dtype('complex64')


def best_of(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = f()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, out


print(f'n = {n}, depth {depth}: {len(gates)} gates')
ms, simp = best_of(lambda: sim._simplify_runs(gates, True, 1e-8, {}))
print(f'simplify                 {ms:8.2f} ms -> {len(simp)} gates')
for name, kw in (('fused_4', dict(compress=4, blocked=False)), ('fused_5', dict(compress=5, blocked=False)),
                 ('blocked quick', dict(compress=5, blocked=dict(tries=8, fusion_orders=1))),
                 ('blocked full', dict(compress=5, blocked=True))):
    ms, ops = best_of(lambda: sim._plan_ops(simp, qubits, n, ct, kw['compress'], kw['blocked']), reps=3)
    extra = blocked_stats(ops) if kw['blocked'] else {'gates': len(ops)}
    print(f'{name:24s} {ms:8.2f} ms  modelled device {sim.estimate_ms(ops, n, ct):7.2f} ms  {extra}')
ms, (ops, info) = best_of(lambda: sim.choose_schedule(simp, qubits, n, ct), reps=3)
print(f'choose_schedule          {ms:8.2f} ms  {info}')
