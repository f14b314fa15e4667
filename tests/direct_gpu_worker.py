"""Worker of tests/test_gpu_round4.py (a fresh process: the library reads HQ_BLOCKED_* once).  The cache-blocked
schedule of a depth-16 benchmark-generator circuit, pass by pass, with whatever HQ_BLOCKED_DIRECT / HQ_BLOCKED_GRID the
caller set, against the per-gate kernels on the same initial state.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import emu_boot  # noqa: E402

EMU = emu_boot.maybe_install()
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hybridq_amd import core  # noqa: E402
from hybridq_amd.blocking import plan_blocked  # noqa: E402
from hybridq_amd.circuits import rqc_1q2q  # noqa: E402
from tolerances import circuit_tol  # noqa: E402

core.use_torch_stream()
out = {}
CASES = [('complex64', torch.float32, 15 if EMU else 24, 3), ('complex128', torch.float64, 14 if EMU else 23, 3)]
if not EMU:
    CASES.append(('complex64', torch.float32, 24, 'auto'))  # the planner's own inner fusion (k = 4 gates: fewer eligible passes)
for ct, ft, n, inner in CASES:
    rng = np.random.default_rng(5)
    gates = rqc_1q2q(n, depth=6 if EMU else 16, seed=9)
    ident = {q: n - 1 - q for q in range(n)}
    tb = (13 if ct == 'complex64' else 12) + (os.environ.get('HQ_BLOCKED_BIG') == '1')  # BIG: 128 KiB tiles, 1024 threads
    ops = plan_blocked(gates, ident, n, tile_bits=min(tb, n), low_bits=5 if ct == 'complex64' else 4, complex_type=ct, inner_max=inner)
    base = torch.from_numpy(rng.standard_normal((2, 1 << n))).to(ft).cuda()
    base /= base.norm()
    ref = base.clone()
    for U, qs in gates:
        core.apply_U(ref[0], ref[1], np.ascontiguousarray(U, dtype=ct), [ident[q] for q in reversed(qs)], n)
    got = base.clone()
    kinds, big = [], []
    for op in ops:
        if op[0] == 'G':
            core.apply_U(got[0], got[1], np.ascontiguousarray(op[1], dtype=ct), op[2], n)
        else:
            core.apply_blocked(got[0], got[1], op[1], op[2], n)
            kinds.append(core.last_kernel_desc().endswith('direct'))
            big.append('1024' in core.last_kernel_desc())
    core.sync()
    err = float(((got - ref).abs().max() / ref.abs().max()).item())
    again = base.clone()
    for op in ops:
        if op[0] == 'G':
            core.apply_U(again[0], again[1], np.ascontiguousarray(op[1], dtype=ct), op[2], n)
        else:
            core.apply_blocked(again[0], again[1], op[1], op[2], n)
    core.sync()
    import hashlib
    out[f'{ct} inner_max={inner}'] = {'n': n, 'passes': len(kinds), 'direct_passes': int(sum(kinds)), 'passes_1024_threads': int(sum(big)), 'err_vs_per_gate': err,
               'tol': circuit_tol(gates, gates, complex_type=ct), 'repeatable': bool(torch.equal(got, again)),
               'sha': hashlib.sha256(got.cpu().numpy().tobytes()).hexdigest()[:24]}
print(json.dumps(out))
