"""Live differential test against the REFERENCE ITSELF, in the build container only: when /root/reference and the compiled
reference core (oracle/_ref) are present, the reference's own ``simulate(optimize='evolution-hybridq')`` runs random
circuits under random options in a subprocess (tests/golden/make_golden.py live), and this package's host side -- on the
numpy test double of the device -- must return the same states.  Skipped anywhere else (the GPU boxes have no reference;
the committed fixtures under tests/golden/ are what travels)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
REF_CORE = os.path.join(ROOT, 'oracle', '_ref')


@pytest.mark.skipif(not (os.path.isdir('/root/reference/hybridq') and os.path.exists(os.path.join(REF_CORE, 'hybridq.so'))),
                    reason='needs /root/reference and oracle/_ref (build container only)')
@pytest.mark.parametrize('seed', [1, 4])  # 4: a width-6 layer whose gates commute only to 1e-5 (to_matrix_gate's inner regrouping); seeds 2, 3, 5, 6 pass as well (5: non-unitary gates around the projection)
def test_random_circuits_and_options_against_the_reference(numpy_device, tmp_path, seed):
    from hybridq_amd.simulation import simulate
    out = str(tmp_path / 'live.npz')
    env = dict(os.environ, LD_LIBRARY_PATH=REF_CORE + ':' + os.environ.get('LD_LIBRARY_PATH', ''), PYTHONDONTWRITEBYTECODE='1')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'), 'live', out, str(seed)],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    z = np.load(out, allow_pickle=False)
    assert int(z['n_cases']) == 10
    for i in range(int(z['n_cases'])):
        if f'c{i}_n_gates' not in z.files:  # the reference itself refused this circuit (active qubits changed)
            continue
        gates = [(z[f'c{i}_U{j}'], tuple(int(q) for q in z[f'c{i}_q{j}'])) for j in range(int(z[f'c{i}_n_gates']))]
        n, ctype = int(z[f'c{i}_n']), str(z[f'c{i}_ctype'])
        psi = simulate(gates, initial_state=str(z[f'c{i}_init']), optimize='evolution-hybridq', complex_type=ctype,
                       compress=int(z[f'c{i}_compress']), simplify=bool(z[f'c{i}_simplify']), qubits=list(range(n)))
        # the same gate stream as the reference's driver builds (simplify, compress, to_matrix_gate), one for one
        from hybridq_amd.simulation import _plan_ops, _simplify_runs
        named = [g for g, name in zip(gates, z[f'c{i}_names']) if str(name) != 'I']  # Gate('I') is stripped by name (:289-291)
        circ = _simplify_runs(named, True, 1e-8, {}) if bool(z[f'c{i}_simplify']) else named
        ours = _plan_ops(circ, list(range(n)), n, np.dtype(ctype), int(z[f'c{i}_compress']), False)
        assert len(ours) == int(z[f'c{i}_f_n']), (seed, i)
        from hybridq_amd.fusion import _embed
        for j, (qs, U) in enumerate(ours):
            fq = tuple(int(q) for q in z[f'c{i}_fq{j}'])
            fU = z[f'c{i}_fU{j}']
            if int(z[f'c{i}_compress']) == 0 and tuple(qs) != fq:
                # unfused, the reference still passes every gate through to_matrix_gate, which sorts its qubits
                # (utils.py:419-464); this driver applies the gate as given: the same operator, indices permuted
                assert sorted(qs) == list(fq), (seed, i, j)
                U = _embed(U, qs, list(fq))
            else:
                assert tuple(qs) == fq, (seed, i, j)
            assert np.abs(np.asarray(U) - fU).max() <= (1e-12 if ctype == 'complex128' else 1e-6) * max(1.0, np.abs(fU).max()), (seed, i, j)
        ref = z[f'c{i}_psi']
        assert psi.dtype == ref.dtype and psi.shape == (2,) * n
        # a Projection (device-side functional gate here, host numpy code there) in the middle of the circuit, as a vector; the gate
        # streams with FunctionalGates in them one for one (tests/fn_stream_checks.py, shared with the golden-fixture test)
        from fn_stream_checks import check_functional_case
        assert 'streams' in check_functional_case(z, i, simulate, tag=f'live seed {seed}')
        # <psi| op |psi> of the reference's final state
        from hybridq_amd.simulation import expectation_value
        op = [(z[f'c{i}_opU{j}'], tuple(int(q) for q in z[f'c{i}_opq{j}'])) for j in range(int(z[f'c{i}_op_n']))]
        ev = expectation_value(ref.astype(np.complex128).reshape((2,) * n), op, qubits_order=list(range(n)), complex_type='complex128')
        assert abs(ev - complex(z[f'c{i}_ev'])) < 1e-10 * max(1.0, abs(complex(z[f'c{i}_ev']))), (seed, i, 'expectation_value')
        # the double computes in float64: against the reference's complex128 run that is rounding only, against its
        # complex64 run the reference's own single-precision error (<= 1e-5 over ~100 non-unitary gates)
        tol = 1e-11 if ctype == 'complex128' else 2e-5
        assert np.abs(psi.reshape(-1) - ref).max() / np.abs(ref).max() < tol, (seed, i, ctype)


@pytest.mark.skipif(not (os.path.isdir('/root/reference/hybridq') and os.path.exists(os.path.join(REF_CORE, 'hybridq.so'))),
                    reason='needs /root/reference and oracle/_ref (build container only)')
def test_noisy_circuits_against_the_reference_dm_simulate(numpy_device, tmp_path):
    """BASELINE config 5's front-end: random circuits with depolarizing / dephasing / amplitude-damping noise through the
    reference's hybridq.dm simulate() (subprocess) and through hybridq_amd.dm.simulate on the double: the same rho."""
    from hybridq_amd import dm
    out = str(tmp_path / 'live_dm.npz')
    env = dict(os.environ, LD_LIBRARY_PATH=REF_CORE + ':' + os.environ.get('LD_LIBRARY_PATH', ''), PYTHONDONTWRITEBYTECODE='1')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'), 'live_dm', out, '5'],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    z = np.load(out, allow_pickle=False)
    assert int(z['n_cases']) == 4
    for i in range(int(z['n_cases'])):
        kinds = bytes(z[f'c{i}_kinds']).decode()
        circuit = []
        for j, kind in enumerate(kinds):
            qs = tuple(int(q) for q in z[f'c{i}_q{j}'])
            if kind == 'K':
                circuit.append(dm.Kraus(list(z[f'c{i}_L{j}']), qs, s=z[f'c{i}_s{j}'], right_ops=list(z[f'c{i}_R{j}'])))
            else:
                circuit.append((z[f'c{i}_U{j}'], qs))
        ref = z[f'c{i}_rho']
        for kw in (dict(compress=4), dict(compress=0), {}):
            rho = dm.simulate(circuit, initial_state=str(z[f'c{i}_init']), complex_type='complex128', **kw).reshape(-1)
            assert np.abs(rho - ref).max() / np.abs(ref).max() < 1e-11, (i, str(z[f'c{i}_kind']), kw)
        d = int(round(np.sqrt(ref.size)))
        assert abs(np.trace(ref.reshape(d, d)) - 1) < 1e-9


def _reference(args, cwd):
    env = dict(os.environ, LD_LIBRARY_PATH=REF_CORE + ':' + os.environ.get('LD_LIBRARY_PATH', ''), PYTHONDONTWRITEBYTECODE='1')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'golden', 'make_golden.py')] + args, cwd=cwd, env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir('/root/reference/hybridq'), reason='needs /root/reference (build container only)')
@pytest.mark.parametrize('seed', [1])
def test_qasm_both_ways_against_the_reference(tmp_path, seed):
    """The reference's to_qasm on a random circuit over every named gate it has (parameters, powers, conj / T, MATRIX
    gates, arbitrary integer labels) -> this package's from_qasm: every matrix and qubit tuple as the reference's
    gate.matrix() / gate.qubits; then this package's to_qasm -> the reference's from_qasm: the same matrices again."""
    from hybridq_amd.qasm import from_qasm, to_qasm
    out = str(tmp_path / 'w.npz')
    _reference(['live_qasm_write', out, str(seed)], str(tmp_path))
    z = np.load(out, allow_pickle=False)
    text = bytes(z['text']).decode()
    gates = from_qasm(text)
    assert len(gates) == int(z['n_gates'])
    seen = set()
    for i, (U, qs) in enumerate(gates):
        assert tuple(qs) == tuple(int(q) for q in z[f'q{i}']), (i, str(z[f'name{i}']))
        assert np.abs(U - z[f'U{i}']).max() < 1e-10 * max(1.0, np.abs(z[f'U{i}']).max()), (i, str(z[f'name{i}']))
        seen.add(str(z[f'name{i}']))
    assert len(seen) >= 23  # every named gate of hybridq/gate/gate.py:127-350 plus MATRIX
    back = str(tmp_path / 'ours.qasm')
    with open(back, 'w') as f:
        f.write(to_qasm(gates))
    out2 = str(tmp_path / 'r.npz')
    _reference(['live_qasm_read', back, out2], str(tmp_path))
    r = np.load(out2, allow_pickle=False)
    assert int(r['n_gates']) == len(gates)
    for i, (U, qs) in enumerate(gates):
        assert [str(q) for q in qs] == [str(q) for q in r[f'q{i}']], i
        assert np.abs(U - r[f'U{i}']).max() < 1e-10 * max(1.0, np.abs(U).max()), i


@pytest.mark.skipif(not (os.path.isdir('/root/reference/hybridq') and os.path.exists(os.path.join(REF_CORE, 'hybridq.so'))),
                    reason='needs /root/reference and oracle/_ref (build container only)')
@pytest.mark.parametrize('seed', [18, 24])  # seeds whose max_n_qubits_matrix = 4 cases told fuse() apart from the reference once
def test_named_gate_circuits_simplify_and_compress_options(numpy_device, tmp_path, seed):
    """Where simplification and fusion have something to decide: circuits of named gates with commuting diagonal gates,
    planted inverse pairs and identities, under random option dictionaries -- fusion.simplify and fusion.fuse return the
    reference's gate lists one for one, and simulate() with those dictionaries the reference's state."""
    from hybridq_amd.fusion import fuse, simplify
    from hybridq_amd.simulation import simulate
    out = str(tmp_path / 'named.npz')
    _reference(['live_named', out, str(seed)], str(tmp_path))
    z = np.load(out, allow_pickle=False)
    assert int(z['n_cases']) == 6
    n = 12
    for i in range(int(z['n_cases'])):
        gates = [(z[f'c{i}_U{j}'], tuple(int(q) for q in z[f'c{i}_q{j}'])) for j in range(int(z[f'c{i}_n_gates']))]
        named = [g for g, name in zip(gates, z[f'c{i}_names']) if str(name) != 'I']
        simp = {'use_matrix_commutation': bool(z[f'c{i}_simp_umc'])}
        if int(z[f'c{i}_simp_mnm']) >= 0:
            simp['max_n_qubits_matrix'] = int(z[f'c{i}_simp_mnm'])
        comp = {'max_n_qubits': int(z[f'c{i}_comp_n']), 'use_matrix_commutation': bool(z[f'c{i}_comp_umc']),
                'max_n_qubits_matrix': int(z[f'c{i}_comp_mnm'])}
        if len(z[f'c{i}_comp_excl']):
            comp['exclude_qubits'] = [int(q) for q in z[f'c{i}_comp_excl']]
        s_ours = simplify(named, atol=1e-8, remove_id_gates=True, **simp)
        assert len(s_ours) == int(z[f'c{i}_s_n']), (seed, i, 'simplify', simp)
        for j, (U, qs) in enumerate(s_ours):
            assert tuple(qs) == tuple(int(q) for q in z[f'c{i}_sq{j}']), (seed, i, j, 'simplify')
            assert np.abs(np.asarray(U) - z[f'c{i}_sU{j}']).max() < 1e-12, (seed, i, j, 'simplify')
        f_ours = fuse(s_ours, comp['max_n_qubits'], complex_type='complex128', **{k: v for k, v in comp.items() if k != 'max_n_qubits'})
        assert len(f_ours) == int(z[f'c{i}_f_n']), (seed, i, 'compress', comp)
        for j, (U, qs) in enumerate(f_ours):
            assert tuple(qs) == tuple(int(q) for q in z[f'c{i}_fq{j}']), (seed, i, j, 'compress')
            assert np.abs(np.asarray(U) - z[f'c{i}_fU{j}']).max() < 1e-12, (seed, i, j, 'compress')
        psi = simulate(named, initial_state=str(z[f'c{i}_init']), optimize='evolution-hybridq', complex_type='complex128',
                       compress=comp, simplify=simp, qubits=list(range(n)))
        ref = z[f'c{i}_psi']
        assert np.abs(psi.reshape(-1) - ref).max() / np.abs(ref).max() < 1e-11, (seed, i)


@pytest.mark.skipif(not (os.path.isdir('/root/reference/hybridq') and os.path.exists(os.path.join(REF_CORE, 'hybridq.so'))),
                    reason='needs /root/reference and oracle/_ref (build container only)')
def test_reference_gate_objects_through_this_driver(tmp_path):
    """The drop-in claim for Python users, literally: circuits built from the REFERENCE'S OWN gate objects (named gates with
    parameters / powers / conj / T, MATRIX gates, string and tuple labels, TupleGates, a StochasticGate under sampling seeds,
    Projection / Measure FunctionalGates, zero-qubit MessageGates, a noisy SuperCircuit of KrausSuperGates) handed unchanged to
    hybridq_amd.simulation.simulate / hybridq_amd.dm.simulate give the states hybridq's own simulate() functions give (tests/reference_objects_worker.py, one process importing both)."""
    env = dict(os.environ, LD_LIBRARY_PATH=REF_CORE + ':' + os.environ.get('LD_LIBRARY_PATH', ''), PYTHONDONTWRITEBYTECODE='1')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'reference_objects_worker.py')], cwd=str(tmp_path), env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'ALL OK' in res.stdout, (res.stdout[-1500:], res.stderr[-2500:])
    for line in ('named / matrix / powers: ok', 'string / tuple labels: ok', 'tuple / stochastic gates: ok',
                 'reference FunctionalGates (Projection, Message, Measure): ok', 'reference FunctionalGates calling the reference dot(): ok',
                 'reference SuperCircuit through dm.simulate: ok'):
        assert line in res.stdout
