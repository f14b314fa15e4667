"""GPU: the HIP library against the golden vectors recorded from the reference."""
import numpy as np
import pytest

import golden_util as gu
from tolerances import C_STRUCTURED, circuit_tol

pytestmark = pytest.mark.gpu


def _tol(dt):
    return 1e-6 if np.dtype(dt) in (np.dtype('float32'), np.dtype('complex64')) else 1e-12


def _trace_widths(z, prefix):
    """Widths of the apply_U calls of a recorded C-ABI trace (swaps move data exactly)."""
    return [len(pos) for kind, pos, _ in gu.trace(z, prefix) if kind == 'U']


def test_apply_U_vs_reference_vectors(torch_cuda):
    from hybridq_amd import core
    torch = torch_cuda
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    for inp, out, U, pos in gu.apply_cases():
        re, im = torch.from_numpy(inp[0].copy()).cuda(), torch.from_numpy(inp[1].copy()).cuda()
        core.apply_U(re, im, U, pos)
        core.sync()
        got = np.stack([re.cpu().numpy(), im.cpu().numpy()])
        err = np.abs(got - out).max() / np.abs(out).max()
        assert err < _tol(inp.dtype) * max(1, len(U) // 8), (list(pos), err, core.last_kernel_desc())


def test_swap_vs_reference_vectors(torch_cuda):
    from hybridq_amd import core
    torch = torch_cuda
    for n, pos, out in gu.swap_cases():
        for dt in (torch.float32, torch.float64, torch.int32, torch.int64):
            a = torch.arange(1 << n).to(dt).cuda()
            core.swap(a, pos)
            core.sync()
            assert (a.cpu().numpy() == out.astype(a.cpu().numpy().dtype)).all(), (dt, list(pos))


def _replay_on_gpu(torch, z, prefix, n, tdt):
    from hybridq_amd import core
    core.set_stream(torch.cuda.current_stream().cuda_stream)
    planes = torch.zeros((2, 1 << n), dtype=tdt, device='cuda')
    core.init_state(planes[0], planes[1], 'basis', 0)
    gu.replay(z, prefix, n, lambda U, pos: core.apply_U(planes[0], planes[1], U, pos, n),
              lambda pos: (core.swap(planes[0], pos, n), core.swap(planes[1], pos, n)))
    core.sync()
    p = planes.cpu().numpy()
    return p[0] + 1j * p[1]


def test_simple_qasm_trace_and_circuit(torch_cuda):
    """BASELINE cfg1 on the GPU: (a) the reference's exact C-ABI call sequence (swaps and
    fused k=4 gates) replayed on device buffers, (b) the 99 unfused gates through
    hybridq_amd.simulate (no swaps)."""
    from hybridq_amd.simulation import simulate
    torch = torch_cuda
    z = gu.load('e2e_simple_qasm.npz')
    n = int(z['n_qubits'])
    stride = int(z['sample_stride'])
    scale = np.abs(z['psi_sample']).max()
    psi = _replay_on_gpu(torch, z, 'trace_', n, torch.float32)
    calls = _trace_widths(z, 'trace_')  # the same 13 fused calls on both sides, both float32
    tol = circuit_tol(calls, calls, c=C_STRUCTURED)
    assert np.abs(psi[::stride] - z['psi_sample']).max() / scale < tol
    assert np.abs(psi[:8] - z['psi_head']).max() / scale < tol
    gates = gu.simple_qasm_gates(z)
    psi2 = simulate(gates, initial_state='0' * n, complex_type='complex64', compress=4).reshape(-1)  # the reference's schedule
    assert np.abs(psi2[::stride] - z['psi_sample']).max() / scale < tol
    # the schedule simulate() picks by itself (99 gates at n = 24: gate by gate) against the reference's 13 fused calls
    psi3 = simulate(gates, initial_state='0' * n, complex_type='complex64').reshape(-1)
    assert np.abs(psi3[::stride] - z['psi_sample']).max() / scale < circuit_tol(gates, calls, c=C_STRUCTURED)
    # |<psi|psi> - 1| <= 2 x the state's own rounding bound (first order in the error)
    assert abs(float(np.vdot(psi2.astype(np.complex128), psi2.astype(np.complex128)).real) - 1.0) < 2 * circuit_tol(calls)


@pytest.mark.parametrize('tag,ct', [('a', 'complex64'), ('b', 'complex128')])
def test_reference_rqc(torch_cuda, tag, ct):
    from hybridq_amd.simulation import simulate
    torch = torch_cuda
    z = gu.load('e2e_rqc.npz')
    n = int(z['n_qubits'])
    exp = z[f'{tag}_psi']
    calls = _trace_widths(z, f'{tag}_trace_')  # what the reference executed (fused), in `ct`
    tol = circuit_tol(calls, calls, complex_type=ct)
    psi = simulate(gu.rqc_gates(z, tag), initial_state='0' * n, complex_type=ct, qubits=list(range(n))).reshape(-1)
    assert np.abs(psi - exp).max() / np.abs(exp).max() < tol
    tdt = torch.float32 if ct == 'complex64' else torch.float64
    psi_r = _replay_on_gpu(torch, z, f'{tag}_trace_', n, tdt)
    assert np.abs(psi_r - exp).max() / np.abs(exp).max() < tol


def test_dm_trace(torch_cuda):
    torch = torch_cuda
    z = gu.load('e2e_dm.npz')
    n = int(z['n_qubits'])
    rho = _replay_on_gpu(torch, z, 'trace_', n, torch.float32)
    exp = z['rho']
    calls = _trace_widths(z, 'trace_')
    assert np.abs(rho - exp).max() / np.abs(exp).max() < circuit_tol(calls, calls)


def test_dm_front_end(torch_cuda):
    """BASELINE cfg5 path: noisy 6-qubit circuit -> hybridq_amd.dm.simulate (12-qubit state
    vector, fused non-unitary gates on the matrix cores) == the reference's rho."""
    from hybridq_amd.dm import Kraus, simulate
    z = gu.load('e2e_dm_circuit.npz')
    kinds = bytes(z['kinds']).decode()
    circuit = []
    for i, kind in enumerate(kinds):
        qs = tuple(int(q) for q in z[f'q{i}'])
        circuit.append(Kraus(list(z[f'L{i}']), qs, s=z[f's{i}']) if kind == 'K' else (z[f'U{i}'], qs))
    n = int(z['n_qubits'])
    from hybridq_amd.dm import to_statevector_circuit
    sv = to_statevector_circuit(circuit)  # the 2n-qubit gates actually applied (z['rho'] is a float32 result)
    tol = circuit_tol(sv, sv)
    for compress in (4, 0):
        rho = simulate(circuit, initial_state='0', complex_type='complex64', compress=compress).reshape(-1)
        assert np.abs(rho - z['rho']).max() / np.abs(z['rho']).max() < tol
    r = rho.reshape(1 << n, 1 << n)
    scale = np.abs(r).max()
    assert abs(np.trace(r).real - 1) < (1 << n) * scale * circuit_tol(sv) and np.abs(r - r.conj().T).max() < 2 * scale * circuit_tol(sv)
    rho128 = simulate(circuit, initial_state='0' * n, complex_type='complex128').reshape(-1)
    assert np.abs(rho128 - z['rho']).max() / np.abs(z['rho']).max() < circuit_tol(sv)


def test_api_simulate_mixed_initial_state(torch_cuda):
    """Reference simulate() with a mixed '01+-' initial state and 150 non-unitary gates:
    complex64/compress=4 and complex128/compress=8 (e2e_api.npz)."""
    from hybridq_amd.simulation import simulate
    z = gu.load('e2e_api.npz')
    gates = gu.rqc_gates(z, 'sim')
    init = str(z['sim_init'])
    n = len(init)
    scale = np.abs(z['sim_psi128']).max()
    p128 = simulate(gates, initial_state=init, complex_type='complex128', compress=8, qubits=list(range(n)))
    assert np.abs(p128.reshape(-1) - z['sim_psi128']).max() / scale < 1e-12
    for kw in (dict(compress=4), dict(compress=0), dict(compress=6), dict(blocked=True)):
        p64 = simulate(gates, initial_state=init, complex_type='complex64', qubits=list(range(n)), **kw)
        assert p64.dtype == np.complex64
        assert np.abs(p64.reshape(-1) - z['sim_psi64']).max() / scale < circuit_tol(gates, gates), kw
        assert np.abs(p64.reshape(-1) - z['sim_psi128']).max() / scale < circuit_tol(gates), kw


def test_api_projection_and_measure(torch_cuda):
    """Device-side Projection / Measure == the reference's functional gates on the same raw
    state with arbitrary integer qubit labels (values recorded from hybridq.gate.Projection /
    Measure; the sampled outcome follows numpy's global RNG exactly like the reference)."""
    from hybridq_amd.functional import Measure, Projection
    from hybridq_amd.simulation import EvolutionState, simulate
    z = gu.load('e2e_api.npz')
    order = [int(q) for q in z['fg_order']]
    n = len(order)
    r = z['fg_state'].reshape((2,) * n)
    # axis a of `r` carries label order[a]; EvolutionState wants sorted labels on the axes
    srt = sorted(order)
    to_sorted = [order.index(q) for q in srt]
    back = [srt.index(q) for q in order]

    def state_of(arr):
        return EvolutionState(srt, complex_type='complex64', initial_state=np.transpose(arr, to_sorted).reshape(-1))

    def result(st):
        return np.transpose(st.to_complex().cpu().numpy().reshape((2,) * n), back).reshape(-1)

    pq = [int(q) for q in z['proj_qubits']]
    for renorm, key in ((False, 'proj_raw'), (True, 'proj_norm')):
        st = state_of(r)
        Projection(str(z['proj_string']), pq, renormalize=renorm).apply_device(st)
        exp = z[key]
        assert np.abs(result(st) - exp).max() / np.abs(exp).max() < 2e-6, key
    mq = [int(q) for q in z['meas_qubits']]
    rn = r / np.linalg.norm(r.reshape(-1))
    st = state_of(rn)
    probs = Measure(mq).probabilities(st)
    assert np.abs(probs - z['meas_probs']).max() < 1e-6
    for j in range(3):
        st = state_of(rn)
        np.random.seed(int(z[f'meas_seed{j}']))
        m = Measure(mq)
        m.apply_device(st)
        exp = z[f'meas_state{j}']
        assert np.abs(result(st) - exp).max() / np.abs(exp).max() < 2e-6, (j, m.outcome)
    # a Projection in the middle of a circuit run by simulate()
    n2 = 12
    g1, g2 = gu.rqc_gates(z, 'fs1'), gu.rqc_gates(z, 'fs2')
    P = Projection(str(z['fs_proj_string']), [int(q) for q in z['fs_proj_qubits']])
    for kw in (dict(compress=4), dict(blocked=True)):
        psi = simulate(g1 + [P] + g2, initial_state='0' * n2, complex_type='complex64', qubits=list(range(n2)), **kw)
        assert np.abs(psi.reshape(-1) - z['fs_psi']).max() / np.abs(z['fs_psi']).max() < circuit_tol(g1 + g2, g1 + g2), kw


def test_api_expectation_value(torch_cuda):
    from hybridq_amd.simulation import expectation_value, simulate
    z = gu.load('e2e_api.npz')
    gates, op = gu.rqc_gates(z, 'ev'), gu.rqc_gates(z, 'evop')
    n = 12
    psi = simulate(gates, initial_state='+' * n, complex_type='complex64', qubits=list(range(n)))
    assert np.abs(psi.reshape(-1) - z['ev_state']).max() / np.abs(z['ev_state']).max() < circuit_tol(gates, gates)
    v = expectation_value(z['ev_state'].reshape((2,) * n), op, qubits_order=list(range(n)))
    assert abs(v - complex(z['ev_value'])) < 2 * circuit_tol(op, op)
    v = expectation_value(psi, op, qubits_order=list(range(n)), complex_type='complex128')
    assert abs(v - complex(z['ev_value'])) < 2 * circuit_tol(gates, gates)


def test_api_dot_and_transpose(torch_cuda):
    """utils.dot() / utils.transpose() outputs recorded from the reference (compiled core path)."""
    from hybridq_amd.dot import dot
    from hybridq_amd.transpose import transpose
    z = gu.load('e2e_api.npz')
    n = int(z['dot_n'])
    for j in range(int(z['dot_cases'])):
        psi, U, axes = z[f'dot{j}_psi'], z[f'dot{j}_U'], z[f'dot{j}_axes']
        k = len(axes)
        tol = (2e-6 if psi.dtype == np.float32 else 1e-13) * 2**k
        exp = z[f'dot{j}_res']
        scale = np.abs(exp).max()
        res = dot(U, np.reshape(np.array(psi), (2,) * (n + 1)), axes_b=axes, b_as_complex_array=True,
                  raise_if_hcore_fails=True)
        assert np.abs(np.asarray(res).reshape(2, -1) - exp).max() / scale < tol, j
        res_c = dot(U, np.reshape(psi[0] + 1j * psi[1], (2,) * n), axes_b=axes, raise_if_hcore_fails=True)
        assert np.abs(np.asarray(res_c).reshape(-1) - z[f'dot{j}_res_complex']).max() / scale < tol, j
        nsb, tr = dot(U, np.reshape(np.array(psi), (2,) * (n + 1)), axes_b=axes, b_as_complex_array=True,
                      swap_back=False, raise_if_hcore_fails=True)
        nsb = np.asarray(nsb).reshape((2,) + (2,) * n)
        if tr is not None:
            nsb = np.stack([transpose(nsb[0], tr), transpose(nsb[1], tr)])
        assert np.abs(nsb.reshape(2, -1) - exp).max() / scale < tol, j
    for j in range(int(z['tr_cases'])):
        nn = int(z[f'tr{j}_n'])
        a = z[f'tr{j}_a'].reshape((2,) * nn)
        b = transpose(np.array(a), [int(x) for x in z[f'tr{j}_axes']], raise_if_hcore_fails=True)
        assert np.array_equal(np.asarray(b).reshape(-1), z[f'tr{j}_res']), j


def test_c_abi_demo_without_python(torch_cuda, tmp_path):
    """examples/abi_demo.cpp: a C++ program that dlopen()s libhq_hip.so and drives the boundary
    declared in include/hq_hip.h (reference entry point apply_U_float32 + device helpers) with
    no Python and no torch in the process: graph state on 20 qubits, every amplitude checked."""
    import os
    import shutil
    import subprocess
    from hybridq_amd import core
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    exe = str(tmp_path / 'abi_demo')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-I', os.path.join(root, 'include'),
                           os.path.join(root, 'examples', 'abi_demo.cpp'), '-o', exe, '-ldl'])
    out = subprocess.run([exe, core._LIB_PATH], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert 'wrong_amplitudes=0' in out.stdout


def test_functional_gate_streams_against_the_reference_record(torch_cuda):
    """Circuits with FunctionalGates in reference-schedule mode against what the REFERENCE did with them
    (tests/golden/e2e_fn_streams.npz, recorded by `make_golden.py fn_streams` from its simplify / compress / simulate,
    circuit/utils.py:166-208, 583-669, 751-759; simulation.py:436-454): the order of the simplified and of the fused gate lists
    with a Projection, a Measure and a closing Projection in them (which element sits where, qubits, matrices via probe
    products), and the final state of the run with a Projection in the middle AS A VECTOR -- non-unitary gates around the
    renormalising projection included, where the order in which gates slide across it is part of the result."""
    from fn_stream_checks import check_functional_case
    from hybridq_amd.simulation import simulate
    z = gu.load('e2e_fn_streams.npz')
    assert int(z['n_cases']) >= 6
    seen = []
    for i in range(int(z['n_cases'])):
        seen += check_functional_case(z, i, simulate, tag='fixture')
    assert seen.count('streams') == int(z['n_cases']) and seen.count('state') >= 4
    assert any(int(z[f'c{i}_compress']) == 6 and not bool(z[f'c{i}_unitary']) for i in range(int(z['n_cases'])))

