"""Parity AT DEPTH on the BASELINE config-2 circuit (n-qubit random circuit, depth 40, Haar 1q/2q
gates, complex64): the HIP path against the reference's own C++ core driven by the reference
protocol (oracle/_ref when it travelled with the snapshot, else the C port) and against a
complex128 evolution of the same circuit as the truth.

What is asserted:
  (a) on every prefix of the circuit where the REFERENCE's own float32 result is still within half
      the bar of the truth (<= 0.5e-6), the HIP result agrees with the reference to the bar (1e-6);
  (b) on every prefix, and at full depth, the HIP result is as close to the truth as the rounding
      model allows (tests/tolerances.py) and not measurably further from it than the reference is;
  (c) complex128: HIP vs reference <= 1e-12 at full depth.
The fused (compress=4) and cache-blocked schedules are held to the same full-depth statements.
"""
import numpy as np
import pytest

from tolerances import BAR, C_MODEL, circuit_tol, rounding_bound, widths

pytestmark = pytest.mark.gpu

N = 22
DEPTH = 40


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.fixture(scope='module')
def cfg2(torch_cuda):
    import oracle
    from hybridq_amd.circuits import rqc_1q2q
    lib = oracle.load_ref() if oracle.have_ref() else oracle.load_port()
    gates = rqc_1q2q(N, depth=DEPTH, seed=N)
    cps = list(range(60, len(gates), 60)) + [len(gates)]
    ref32, i32 = oracle.evolve_reference_protocol(lib, gates, N, complex_type='complex64', checkpoints=cps)
    ref64, i64 = oracle.evolve_reference_protocol(lib, gates, N, complex_type='complex128', checkpoints=cps)
    i32['checkpoints'][len(gates)] = ref32
    i64['checkpoints'][len(gates)] = ref64
    return dict(lib=lib, gates=gates, cps=cps, ref32=i32['checkpoints'], truth=i64['checkpoints'])


def test_depth40_per_gate_prefixes(cfg2, capsys):
    from hybridq_amd.simulation import EvolutionState
    gates, cps = cfg2['gates'], cfg2['cps']
    st = EvolutionState(list(range(N)), complex_type='complex64', initial_state='0' * N)
    rows, done = [], 0
    for c in cps:
        for U, qs in gates[done:c]:
            st.apply(U, qs)
        done = c
        psi = st.to_complex().cpu().numpy()
        truth, ref = cfg2['truth'][c], cfg2['ref32'][c]
        rows.append((c, _rel(ref, truth), _rel(psi, truth), _rel(psi, ref), rounding_bound(widths(gates[:c]))))
    with capsys.disabled():
        print(f'\n  config-2 generator n={N} depth={DEPTH} (reference core: {cfg2["lib"].kind}); bound uses c={C_MODEL}')
        print('  gates  ref32-vs-f64  hip32-vs-f64  hip32-vs-ref32  model-bound   c_ref   c_hip')
        for c, er, eg, d, b in rows:
            print(f'  {c:5d}  {er:11.3e}  {eg:11.3e}  {d:13.3e}  {b:10.3e}  {er / b * C_MODEL:6.3f}  {eg / b * C_MODEL:6.3f}')
    for c, er, eg, d, b in rows:
        if er <= 0.5 * BAR[np.dtype('complex64')]:  # (a)
            assert d <= BAR[np.dtype('complex64')], (c, er, eg, d)
        assert eg <= max(BAR[np.dtype('complex64')], b), (c, eg, b)  # (b) within the rounding model
        assert er <= max(BAR[np.dtype('complex64')], b), (c, er, b)  # ... which the reference obeys too
        assert d <= circuit_tol(gates[:c], gates[:c]), (c, d)
    # (b) not measurably further from the truth than the reference: the two errors are draws of the
    # same random walk (max over 2^22 amplitudes), equal within ~10 %; a kernel defect is not
    c, er, eg, _, _ = rows[-1]
    assert eg <= 1.15 * er, (eg, er)


@pytest.mark.parametrize('name,kw', [('fused_k4', dict(compress=4)), ('fused_k5', dict(compress=5)),
                                     ('blocked', dict(blocked=True)), ('evolution_hip', dict(optimize='evolution-hip'))])
def test_depth40_schedules_full_depth(cfg2, name, kw, capsys):
    from hybridq_amd.simulation import simulate
    gates = cfg2['gates']
    psi = simulate(gates, initial_state='0' * N, complex_type='complex64', qubits=list(range(N)), simplify=False,
                   **kw).reshape(-1)
    truth, ref = cfg2['truth'][len(gates)], cfg2['ref32'][len(gates)]
    er, eg, d = _rel(ref, truth), _rel(psi, truth), _rel(psi, ref)
    with capsys.disabled():
        print(f'\n  {name}: ref32-vs-f64 {er:.3e}  hip32-vs-f64 {eg:.3e}  hip32-vs-ref32 {d:.3e}')
    # fewer roundings of the state than gate-by-gate: at least as close to the truth as the reference
    assert eg <= 1.05 * er, (name, eg, er)
    assert d <= circuit_tol(gates, gates), (name, d)


def test_depth40_complex128(cfg2):
    from hybridq_amd.simulation import simulate
    gates = cfg2['gates']
    truth = cfg2['truth'][len(gates)]
    for kw in (dict(compress=0), dict(compress=4), dict(blocked=True)):
        psi = simulate(gates, initial_state='0' * N, complex_type='complex128', qubits=list(range(N)), simplify=False,
                       **kw).reshape(-1)
        assert _rel(psi, truth) <= BAR[np.dtype('complex128')], kw  # (c)


def _other_generator(kind):
    """BASELINE configs 4 and 5 at a size the reference core finishes in seconds: (gates, n)."""
    from hybridq_amd.circuits import dense_kq, rqc_1q2q
    if kind == 'config4_dense_k34':  # 200 Haar 3-/4-qubit gates, unfused (bench.py --workload dense_k34)
        return dense_kq(N, n_gates=200, seed=34), N
    # config 5 (bench.py --workload dm): an 11-qubit noisy circuit as a 22-qubit state vector -- every gate U becomes U
    # on the left copy, conj(U) on the right copy, then a depolarizing superoperator (one NON-unitary 2k-qubit gate)
    from hybridq_amd.dm import depolarizing, to_statevector_circuit
    nq = N // 2
    noisy = []
    for U, qs in rqc_1q2q(nq, depth=10, seed=nq):
        noisy.append((U, qs))
        noisy.append(depolarizing(qs, 0.01 if len(qs) == 1 else 0.02))
    sv = to_statevector_circuit(noisy)
    labels = sorted({q for _, qs in sv for q in qs})
    index = {lab: i for i, lab in enumerate(labels)}
    return [(U, tuple(index[q] for q in qs)) for U, qs in sv], 2 * nq


@pytest.mark.parametrize('kind', ['config4_dense_k34', 'config5_noisy_dm'])
def test_other_baseline_generators_at_depth(torch_cuda, kind, capsys):
    """The config-4 and config-5 generators through the same three-way comparison: HIP float32
    against the reference core's float32 run and both against the reference core's complex128 run (the truth)."""
    import oracle
    from hybridq_amd.simulation import simulate
    lib = oracle.load_ref() if oracle.have_ref() else oracle.load_port()
    gates, n = _other_generator(kind)
    ref32, _ = oracle.evolve_reference_protocol(lib, gates, n, complex_type='complex64')
    truth, _ = oracle.evolve_reference_protocol(lib, gates, n, complex_type='complex128')
    bound = rounding_bound(widths(gates))
    er = _rel(ref32, truth)
    lines = []
    for name, kw in (('per_gate', dict(compress=0)), ('fused_k4', dict(compress=4)), ('blocked', dict(blocked=True))):
        psi = simulate(gates, initial_state='0' * n, complex_type='complex64', qubits=list(range(n)), simplify=False, **kw).reshape(-1)
        eg, d = _rel(psi, truth), _rel(psi, ref32)
        lines.append(f'  {kind} {name}: {len(gates)} gates, ref32-vs-f64 {er:.3e} (c = {er / bound * C_MODEL:.2f})  hip32-vs-f64 {eg:.3e} '
                     f'(c = {eg / bound * C_MODEL:.2f})  hip32-vs-ref32 {d:.3e}  literal_bar_met: {d <= BAR[np.dtype("complex64")]}')
        assert eg <= max(BAR[np.dtype('complex64')], bound), (kind, name, eg, bound)
        assert eg <= 1.15 * max(er, 0.5 * BAR[np.dtype('complex64')]), (kind, name, eg, er)  # as close to the truth as the reference
        assert d <= circuit_tol(gates, gates), (kind, name, d)
    psi = simulate(gates, initial_state='0' * n, complex_type='complex128', qubits=list(range(n)), simplify=False, compress=0).reshape(-1)
    e128 = _rel(psi, truth)
    with capsys.disabled():
        print()
        for ln in lines:
            print(ln)
        print(f'  {kind} complex128 per_gate vs reference complex128: {e128:.3e}')
    assert e128 <= BAR[np.dtype('complex128')] * (1 if kind == 'config4_dense_k34' else 4), (kind, e128)  # config 5 is non-unitary (norm shrinks)
