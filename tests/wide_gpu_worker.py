"""Worker of tests/test_gpu_round4.py::test_wide_kernels_both_loop_forms_against_the_oracle (a fresh process: the library reads
HQ_GEMM_PIPE / HQ_BIG_TWOBASE once).  One gate of every width k = 5..10 (role kernel with its operand table in LDS, tile GEMM)
at several position patterns, both precisions, PER CALL against the ORACLE (the reference core, U.h:123-202, on the same
state and matrix).  Allowed: north_star's literal bar (1e-6 / 1e-12), or -- where one call alone exceeds it whatever the
implementation, k >= 7 in float32: two float32 accumulations of 2^(k+1) terms in different orders -- the rounding model of
tests/tolerances.py with c = 1 (the same rule as tests/test_gpu_parity.py: wide_tol); `literal_bar_met` reports the bar itself.
Prints one JSON line with the errors and a digest of every result."""
import hashlib
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
from hybridq_amd.circuits import haar_unitary  # noqa: E402
import oracle  # noqa: E402
from tolerances import circuit_tol  # noqa: E402

ORACLE = oracle.load_ref() if oracle.have_ref() else oracle.load_port()
core.use_torch_stream()
out = {}
for ct, ft, n in (('complex64', torch.float32, 15 if EMU else 18), ('complex128', torch.float64, 14 if EMU else 17)):  # (the sizes of tests/test_gpu_parity.py: the wide tolerance is a tail bound that grows slowly with the number of amplitudes)
    rng = np.random.default_rng(11)
    base = torch.from_numpy(rng.standard_normal((2, 1 << n))).to(ft).cuda()
    base /= base.norm()
    host0 = base.cpu().numpy()
    psi0 = host0[0] + 1j * host0[1]
    for k in range(5, (10 if ct == 'complex64' else 9) + 1):
        patterns = [sorted(int(p) for p in rng.permutation(n)[:k]),               # anywhere
                    sorted(int(p) for p in 2 + rng.permutation(n - 2)[:k]),       # no vector-component bit among the targets
                    list(range(k)), list(range(n - k, n))]                         # lowest / highest
        for pi, pos in enumerate(patterns[:2] if EMU else patterns):
            U = np.ascontiguousarray(haar_unitary(1 << k, rng), dtype=ct)
            qs = [n - 1 - p for p in reversed(pos)]  # identity placement: qubit q sits at index bit n - 1 - q
            exp, _ = oracle.evolve_reference_protocol(ORACLE, [(U, tuple(qs))], n, initial_state=psi0, qubits=list(range(n)), complex_type=ct)
            got = base.clone()
            core.apply_U(got[0], got[1], U, pos, n)
            core.sync()
            g = got.cpu().numpy()
            err = float(np.abs((g[0] + 1j * g[1]) - exp).max() / np.abs(exp).max())
            bar = 1e-6 if ct == 'complex64' else 1e-12
            out[f'{ct} k={k} pattern={pi}'] = {'err_vs_oracle': err, 'bar': bar, 'tol': max(bar, circuit_tol([k], [k], complex_type=ct, c=1.0)),
                                               'literal_bar_met': bool(err <= bar), 'kernel': core.last_kernel_desc(),
                                               'sha': hashlib.sha256(g.tobytes()).hexdigest()[:24], 'oracle': ORACLE.kind}
print(json.dumps(out))
