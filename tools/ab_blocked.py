"""A/B of the cache-blocked schedule of the n = 30 benchmark circuit (HQ_HIP_LIBRARY selects the build).
    python tools/ab_blocked.py [n] [complex64|complex128] [tile bits: 13 / 12 by default; 14 / 13 with HQ_BLOCKED_BIG=1 = 128 KiB tiles]"""
import os
import sys
import time

os.environ.setdefault('OPENBLAS_NUM_THREADS', '8')
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridq_amd import core  # noqa: E402
from hybridq_amd.blocking import blocked_stats, plan_blocked  # noqa: E402
from hybridq_amd.circuits import rqc_1q2q  # noqa: E402
from hybridq_amd.simulation import EvolutionState  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ctype = sys.argv[2] if len(sys.argv) > 2 else 'complex64'
tb = int(sys.argv[3]) if len(sys.argv) > 3 else (13 if ctype == 'complex64' else 12)
gates = rqc_1q2q(n, depth=40, seed=n)
# blocked passes run from torch's allocator (as simulate() and bench.py's blocked leg do: ~7 % faster than the tuned VMM placement)
state = EvolutionState(list(range(n)), complex_type=ctype, initial_state='0' * n, placement=os.environ.get('HQ_AB_PLACEMENT', 'plain'))
mode = sys.argv[4] if len(sys.argv) > 4 else ''
as_json = mode in ('json', 'json_full')  # bench.py: one JSON line ('json': the planner's own fusion only; 'json_full': the `blocked` leg -- planner statistics and the no-fusion schedule as well)
low_bits = int(os.environ.get('HQ_AB_LOW_BITS', 5 if ctype == 'complex64' else 4))
report = {}
for kw in ((dict(),) if mode == 'json' else (dict(), dict(inner_max=0))):
    t_p = time.perf_counter()
    ops = plan_blocked(gates, state.map, n, **{**dict(tile_bits=tb, low_bits=low_bits, complex_type=ctype, seeds=int(os.environ.get('HQ_AB_SEEDS', '4'))), **kw})  # (best of 4 seeds by modelled time: host work before the clock, reported as plan_seconds)
    t_plan = time.perf_counter() - t_p
    packed = [('B', op[1], core.pack_blocked(op[2], ctype)) if op[0] == 'B' else op for op in ops]

    def run():
        for op in packed:
            if op[0] == 'G':
                core.apply_U(state.planes[0], state.planes[1], op[1], op[2], n)
            else:
                core.apply_blocked(state.planes[0], state.planes[1], op[1], packed=op[2], n_qubits=n)
    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3 if not kw or not as_json else 1):
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    st = blocked_stats(ops)
    if as_json and kw:  # the same blocked schedule WITHOUT algebraic fusion: every original gate on its own inside the LDS tiles
        report['no_fusion'] = {'blocked_passes': st['blocked_passes'], 'plain_gates': st['plain_gates'], 'inner_gates': st['inner_gates'], 'ms_per_step': round(ts[0], 3)}
        continue
    if as_json:
        n_direct = 0
        for op in packed:  # how many passes took the direct first gate (the library reports the last launch)
            if op[0] == 'B':
                core.apply_blocked(state.planes[0], state.planes[1], op[1], packed=op[2], n_qubits=n)
                n_direct += core.last_kernel_desc().endswith('direct')
        torch.cuda.synchronize()
        report.update({'tile_bits': tb, 'low_bits': low_bits, 'passes': st['blocked_passes'], 'plain_gates': st['plain_gates'], 'inner_gates': st['inner_gates'],
                       'direct_passes': n_direct, 'kernel': core.last_kernel_desc().split(' tb=')[0], 'ms_per_step': [round(t, 3) for t in ts],
                       'plan_seconds': round(t_plan, 3), 'selfcheck': core.blocked_selfcheck()})
        if mode == 'json_full':
            report['stats'] = dict(st, inner_k_histogram={str(k): v for k, v in st['inner_k_histogram'].items()})
        continue
    print(os.path.basename(os.environ.get('HQ_HIP_LIBRARY', 'in-tree')), f'tb={tb}', core.last_kernel_desc().split('>')[0].split('<')[-1], kw, f"passes {st['blocked_passes']} + {st['plain_gates']} plain, inner {st['inner_gates']}:",
          ' '.join('%.1f' % t for t in ts), 'ms', flush=True)
if as_json:
    import json
    print(json.dumps(report), flush=True)
