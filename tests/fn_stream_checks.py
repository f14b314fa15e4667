"""Shared by tests/test_reference_live_host.py (against the reference run live, build container only) and
tests/test_golden_api_host.py (against tests/golden/e2e_fn_streams.npz, which travels): circuits with FunctionalGates through
this driver's reference-schedule mode -- the projection run compared as a VECTOR, and the gate streams (compress with a
Projection in the middle; simplify + compress with a Projection, a Measure and a closing Projection) one for one: which
element sits where, qubits, matrices.  `z` maps `c{i}_...` keys as tests/golden/make_golden.py: live_cases writes them; a fixture
may hold a matrix as its product with a fixed probe vector (`...Uv{j}` instead of `...U{j}`: 2^k numbers instead of 4^k).
Not collected (no test_ prefix)."""
import numpy as np


def probe(d):
    """The fixed vector a fixture multiplies a d x d matrix with (any mismatch of the matrix shows in the product)."""
    rng = np.random.default_rng(1234 + d)
    return rng.standard_normal(d) + 1j * rng.standard_normal(d)


def check_functional_case(z, i, simulate, tag=''):
    from hybridq_amd.functional import Measure, Projection
    from hybridq_amd.fusion import _embed
    from hybridq_amd.simulation import _plan_ops, _simplify_runs
    pre = f'c{i}_'
    files = set(z.files)
    n, comp = int(z[pre + 'n']), int(z[pre + 'compress'])
    gates = [(np.asarray(z[f'{pre}U{j}'], dtype=np.complex128), tuple(int(q) for q in z[f'{pre}q{j}'])) for j in range(int(z[pre + 'n_gates']))]
    # Gate('I') is stripped by NAME before anything else in the reference (simulation.py:289-291); (U, qubits) pairs carry no
    # names, so the caller drops them -- left in, they would take part in the compression walk
    keep = [j for j, name in enumerate(z[pre + 'names']) if str(name) != 'I']
    named = [gates[j] for j in keep]
    n_before = lambda c: sum(1 for j in keep if j < c)  # noqa: E731
    cut = int(z[pre + 'proj_cut'])
    P1 = Projection(str(z[pre + 'proj_bits']), [int(q) for q in z[pre + 'proj_q']])
    done = []
    if pre + 'proj_psi' in files:
        pp = simulate(named[:n_before(cut)] + [P1] + named[n_before(cut):], initial_state=str(z[pre + 'init']), optimize='evolution-hybridq',
                      complex_type='complex128', compress=comp, simplify=False, qubits=list(range(n))).reshape(-1)
        pref = z[pre + 'proj_psi']
        if not pref.any():  # nothing survived the projection (projection.py:58-66): all zeros on both sides
            assert not pp.any(), (tag, i, 'projection onto nothing')
        else:
            # compared as VECTORS, non-unitary circuits included: the reference's compression slides gates on other qubits across
            # the (renormalising) projection, which changes the norm at the moment of renormalisation; this driver reproduces that
            # walk (fusion.Opaque; circuit/utils.py:630-648), so the states agree, not merely the rays
            assert np.abs(pp - pref).max() / np.abs(pref).max() < 1e-10, (tag, i, 'projection', bool(z[pre + 'unitary']))
        done.append('state')
    if pre + 'pj_f_n' not in files:
        return done
    fns = [P1, Measure([int(q) for q in z[pre + 'fn_mq']]), Projection('1', [int(q) for q in z[pre + 'fn_p2q']])]
    c3 = int(z[pre + 'n_gates']) // 3

    def check_stream(ops, stag, kind):
        assert len(ops) == int(z[f'{stag}_{kind}_n']), (tag, i, stag, kind, len(ops), int(z[f'{stag}_{kind}_n']))
        for j, op in enumerate(ops):
            want = int(z[f'{stag}_{kind}F{j}'])
            if want >= 0:
                assert op is fns[want], (tag, i, stag, kind, j)
                continue
            assert not any(op is f for f in fns), (tag, i, stag, kind, j)
            qs, U = op if kind == 'f' else (op[1], op[0])
            fq = tuple(int(q) for q in z[f'{stag}_{kind}q{j}'])
            if tuple(qs) != fq:  # unfused gates keep the order they were given in; the reference sorts (to_matrix_gate)
                assert kind == 's' or comp == 0, (tag, i, stag, kind, j)
                assert sorted(qs) == sorted(fq), (tag, i, stag, kind, j)
                U = _embed(U, qs, list(fq))
            U = np.asarray(U)
            if f'{stag}_{kind}U{j}' in files:
                fU = z[f'{stag}_{kind}U{j}']
                assert np.abs(U - fU).max() <= 1e-12 * max(1.0, np.abs(fU).max()), (tag, i, stag, kind, j)
            else:
                fv = z[f'{stag}_{kind}Uv{j}']
                assert np.abs(U @ probe(U.shape[0]) - fv).max() <= 1e-11 * max(1.0, np.abs(fv).max()), (tag, i, stag, kind, j)

    with_p = named[:n_before(cut)] + [fns[0]] + named[n_before(cut):]
    check_stream(_plan_ops(with_p, list(range(n)), n, np.dtype('complex128'), comp, False), pre + 'pj', 'f')
    with_all = named[:n_before(c3)] + [fns[0]] + named[n_before(c3):n_before(2 * c3)] + [fns[1]] + named[n_before(2 * c3):] + [fns[2]]
    simp = _simplify_runs(with_all, True, 1e-8, {})
    check_stream(simp, pre + 'fn', 's')
    check_stream(_plan_ops(simp, list(range(n)), n, np.dtype('complex128'), comp, False), pre + 'fn', 'f')
    return done + ['streams']
