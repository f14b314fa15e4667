"""Worker of tests/test_gpu_round4.py (a fresh process: the library reads HQ_BLOCKED_* once).  The cache-blocked
schedule of a depth-16 benchmark-generator circuit, pass by pass, with whatever HQ_BLOCKED_* the caller set, against the
ORACLE (the reference core driven by the reference protocol on the same gates and the same initial state; U.h:87-95) --
the parity statement -- and, as extras, against the per-gate HIP kernels and against a second run.  Prints one JSON line."""
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
import oracle  # noqa: E402

ORACLE = oracle.load_ref() if oracle.have_ref() else oracle.load_port()

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
    host0 = base.cpu().numpy()
    exp, _ = oracle.evolve_reference_protocol(ORACLE, gates, n, initial_state=host0[0] + 1j * host0[1], qubits=list(range(n)), complex_type=ct)
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
    got_h = got.cpu().numpy()
    err_oracle = float(np.abs((got_h[0] + 1j * got_h[1]) - exp).max() / np.abs(exp).max())
    again = base.clone()
    for op in ops:
        if op[0] == 'G':
            core.apply_U(again[0], again[1], np.ascontiguousarray(op[1], dtype=ct), op[2], n)
        else:
            core.apply_blocked(again[0], again[1], op[1], op[2], n)
    core.sync()
    import hashlib
    out[f'{ct} inner_max={inner}'] = {'n': n, 'passes': len(kinds), 'direct_passes': int(sum(kinds)), 'passes_1024_threads': int(sum(big)), 'err_vs_per_gate': err,
               'err_vs_oracle': err_oracle, 'oracle': ORACLE.kind, 'literal_bar_met': bool(err_oracle <= (1e-6 if ct == 'complex64' else 1e-12)),
               'tol': circuit_tol(gates, gates, complex_type=ct), 'repeatable': bool(torch.equal(got, again)),
               'sha': hashlib.sha256(got.cpu().numpy().tobytes()).hexdigest()[:24]}
out['selfcheck'] = core.blocked_selfcheck()
print(json.dumps(out))
