"""Tolerances of the end-to-end parity tests, computed from the circuit instead of chosen.

north_star's bar is max|d| / max|psi| <= 1e-6 (complex64) / 1e-12 (complex128) against the
reference's evolution path.  A single ``apply_U`` call meets it with a wide margin (every per-call
test uses the bar itself).  Over a whole circuit BOTH float32 implementations round the state once
per gate, and each of the 2^(k+1) real fused-multiply-adds behind an output amplitude of a k-qubit
gate (U.h:87-95: ``re += Ur*xr - Ui*xi``; here the k-ordered MFMA/fma chain) rounds once, so the
distance of either one to the exact evolution grows like a random walk:

    err  ~  c * u * sqrt( sum_gates 2^(k_g + 1) ),      u = eps/2 (unit round-off)

Measured constants (tests/test_gpu_depth_parity.py prints them on every run): BASELINE config-2
generator, n = 22..24, depth 40 (660-720 one- and two-qubit gates): reference-f32 vs complex128
c = 0.16..0.40 along the circuit, this repo's per-gate path the same 0.16..0.40, fused 0.28, blocked
0.22; the 13 fused four-qubit gates of examples/circuit_simple.qasm (structured H/CZ/T products):
0.73 between two float32 runs (0.52 per evolution); config-4 generator (200 dense 3-/4-qubit gates)
0.31..0.34 (reference 0.32); config-5 generator (noisy 11-qubit circuit as a 22-qubit state vector,
non-unitary superoperators) 0.11..0.25 (reference 0.27).  The tests take c = 0.5 (round 4; 0.6 in round 3, 1.0 before: VERDICT r02 / r03 -- 1.25x above the
largest constant measured, so a kernel that loses a factor of two in accuracy fails, rounding noise does
not) and never go below the bar itself.  Every end-to-end check also reports whether the LITERAL bar was
met (`literal_bar_met`), whatever the model allows.  Two independent float32 evolutions may be apart by the
root-sum-square of their individual bounds.  The depth test does NOT lean on this bound alone: it
asserts that the HIP result is as close to the complex128 truth as the reference's own float32
result is, prefix by prefix.
"""
import numpy as np

BAR = {np.dtype('complex64'): 1e-6, np.dtype('complex128'): 1e-12}
_UNIT = {np.dtype('complex64'): float(np.finfo(np.float32).eps) / 2, np.dtype('complex128'): float(np.finfo(np.float64).eps) / 2}
C_MODEL = 0.5
#: circuits of STRUCTURED gates (examples/circuit_simple.qasm: H / CZ / T / sqrt-X products fused to 4 qubits; matrix
#: entries 0, +-1/2, +-1/sqrt(2)...): the rounding errors of successive gates are correlated instead of a random walk;
#: measured 0.73 between the reference's float32 run and ours (both schedules), so these tests take 0.8
C_STRUCTURED = 0.8
#: NON-unitary gates (the reference's own tests use Ginibre matrices, tests.py:299-391): the kappa weighting below is a
#: geometric-mean heuristic between "no amplification" and the worst case; measured constants against it 0.41 (n = 22) and
#: 0.73 (n = 20, 600 Ginibre 1-/2-qubit gates fused to 4 qubits, median kappa 5, max 300), so these terms keep c = 1
C_NONUNITARY = 1.0


def widths(gates):
    """Per gate (k, kappa): width and 2-norm condition number (1 for unitaries and for entries
    given as bare widths, e.g. the calls of a recorded trace)."""
    out = []
    for g in gates:
        if isinstance(g, (tuple, list)) and len(g) == 2 and not np.isscalar(g[1]):
            U, k = np.asarray(g[0]), len(g[1])
            kappa = 1.0
            if U.ndim == 2 and U.shape[0] == U.shape[1]:
                sv = np.linalg.svd(U.astype(np.complex128), compute_uv=False)
                kappa = float(sv[0] / max(sv[-1], 1e-300))
            out.append((k, kappa if kappa > 1.0 + 1e-6 else 1.0))
        elif isinstance(g, (tuple, list)) and len(g) == 2:
            out.append((int(g[0]), float(g[1])))
        else:
            out.append((int(g), 1.0))
    return out


def rounding_bound(gate_widths, complex_type='complex64', c=None):
    """c * u * sqrt(sum kappa_g 2^(k_g+1)): modelled distance of ONE evolution in `complex_type` from
    the exact one.  kappa_g = 1 for unitary gates.  A NON-unitary gate (the reference's own tests
    use Ginibre matrices, tests.py:299-391, 2335-2369) can amplify the relative error already
    present by up to its condition number (worst case kappa^2 in this sum, nothing for a typical
    direction); the model takes the geometric mean, kappa.  Measured on 600 Ginibre 1-/2-qubit
    gates (median kappa 5, max 300, n = 22): 4.4e-6 against a bound of 1.1e-5 (c = 0.41, the same
    constant as for unitary circuits)."""
    ct = np.dtype(complex_type)
    tot_u = tot_n = 0.0
    for w in gate_widths:
        k, kappa = (w if isinstance(w, tuple) else (w, 1.0))
        if kappa > 1.0:
            tot_n += kappa * 2.0 ** (k + 1)
        else:
            tot_u += 2.0 ** (k + 1)
    cu = C_MODEL if c is None else c
    cn = max(cu, C_NONUNITARY)
    return _UNIT[ct] * float(np.sqrt(cu * cu * tot_u + cn * cn * tot_n))


def circuit_tol(gates_a, gates_b=None, complex_type='complex64', c=None):
    """Allowed max|d| / max|psi| between an evolution applying `gates_a` and one applying `gates_b`
    (both in `complex_type`; ``gates_b=None``: the other side is exact / higher precision).
    Never below north_star's bar."""
    ct = np.dtype(complex_type)
    ea = rounding_bound(widths(gates_a), ct, c)
    eb = rounding_bound(widths(gates_b), ct, c) if gates_b is not None else 0.0
    return max(BAR[ct], float(np.hypot(ea, eb)))
