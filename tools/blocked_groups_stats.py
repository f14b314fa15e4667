"""How many workgroup barriers the barrier-free groups of round 4 remove from the cache-blocked plan of the benchmark circuit.
No GPU needed: every pass of the n = 30 plan is replayed on a one-tile state (its tile bits renumbered 0..12) through the
host EMULATION of the library (tests/emu), whose apply_blocked entry reports `barriers=` in last_kernel_desc -- the
grouping itself is host code of the product (hq_apply.hip).
    python tools/blocked_groups_stats.py [n] [depth]
With HQ_BLOCKED_DIRECT=1 in the environment the same replay also counts the passes that take apply_blocked_direct_kernel
(a k <= 3 matrix-core gate that may run first and whose register digits lie above tile-local vector bit 2)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import emu_util  # noqa: E402
from hybridq_amd.blocking import plan_blocked  # noqa: E402
from hybridq_amd.circuits import rqc_1q2q  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 40
core = emu_util.emu_core()
gates = rqc_1q2q(n, depth=depth, seed=n)
ops = plan_blocked(gates, {q: n - 1 - q for q in range(n)}, n)
tb = 13
re, im, free = emu_util.device_planes(core, tb, np.float32)
tot_g = tot_b = passes = n_direct = 0
hist = {}
for op in ops:
    if op[0] != 'B':
        continue
    local = {int(p): i for i, p in enumerate(op[1])}
    inner = [(U, [local[int(p)] for p in pos]) for U, pos in op[2]]
    re[:] = 0
    im[:] = 0
    re[0] = 1
    core.apply_blocked(re, im, np.arange(tb, dtype=np.uint32), inner, tb)
    d = dict(kv.split('=') for kv in core.last_kernel_desc().split() if '=' in kv)
    g, b = int(d['gates']), int(d['barriers'])
    tot_g += g
    tot_b += b
    passes += 1
    n_direct += core.last_kernel_desc().endswith('direct')
    hist[(g, b)] = hist.get((g, b), 0) + 1
free()
print(f'n = {n}, depth {depth}: {passes} blocked passes, {tot_g} inner gates, {tot_b} workgroup barriers after gates '
      f'({tot_g - tot_b} removed = {100.0 * (tot_g - tot_b) / max(1, tot_g):.0f} %)')
print('  (gates, barriers) per pass:', dict(sorted(hist.items())))
if os.environ.get('HQ_BLOCKED_DIRECT') == '1':
    print(f'  passes with the tile movement folded into their first gate (HQ_BLOCKED_DIRECT=1): {n_direct} of {passes}')
