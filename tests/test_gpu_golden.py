"""GPU: the HIP library against the golden vectors recorded from the reference."""
import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


def _tol(dt):
    return 1e-6 if np.dtype(dt) in (np.dtype('float32'), np.dtype('complex64')) else 1e-12


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
    assert np.abs(psi[::stride] - z['psi_sample']).max() / scale < 5e-6
    assert np.abs(psi[:8] - z['psi_head']).max() / scale < 5e-6
    psi2 = simulate(gu.simple_qasm_gates(z), initial_state='0' * n, complex_type='complex64').reshape(-1)
    assert np.abs(psi2[::stride] - z['psi_sample']).max() / scale < 1e-5
    assert abs(float(np.vdot(psi2.astype(np.complex128), psi2.astype(np.complex128)).real) - 1.0) < 1e-5


@pytest.mark.parametrize('tag,ct', [('a', 'complex64'), ('b', 'complex128')])
def test_reference_rqc(torch_cuda, tag, ct):
    from hybridq_amd.simulation import simulate
    torch = torch_cuda
    z = gu.load('e2e_rqc.npz')
    n = int(z['n_qubits'])
    exp = z[f'{tag}_psi']
    tol = 5e-6 if ct == 'complex64' else 1e-12
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
    assert np.abs(rho - exp).max() / np.abs(exp).max() < 5e-6


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
    for compress in (4, 0):
        rho = simulate(circuit, initial_state='0', complex_type='complex64', compress=compress).reshape(-1)
        assert np.abs(rho - z['rho']).max() / np.abs(z['rho']).max() < 5e-6
    r = rho.reshape(1 << n, 1 << n)
    assert abs(np.trace(r).real - 1) < 1e-5 and np.abs(r - r.conj().T).max() < 1e-6
    rho128 = simulate(circuit, initial_state='0' * n, complex_type='complex128').reshape(-1)
    assert np.abs(rho128 - z['rho']).max() / np.abs(z['rho']).max() < 5e-6
