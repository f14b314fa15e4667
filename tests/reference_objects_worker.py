"""Worker of tests/test_reference_live_host.py::test_reference_gate_objects_through_this_driver (build container only).

Imports the REFERENCE (with make_golden's stand-ins for its three absent third-party modules), builds circuits out of its
own gate objects -- named gates, MATRIX gates, powers / conj / T, TupleGates, a StochasticGate, Projection and Measure
FunctionalGates, zero-qubit MessageGates, string and tuple qubit labels -- and hands the SAME objects to
hybridq.circuit.simulation.simulate and to hybridq_amd.simulation.simulate (host side for real, device replaced by the numpy
test double).  Prints one line per scenario and 'ALL OK'."""
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden  # noqa: E402

make_golden.install_stubs()
sys.path.insert(0, make_golden.REF)
import hybridq.circuit.simulation.simulation as refsim  # noqa: E402
from hybridq.circuit import Circuit  # noqa: E402
from hybridq.circuit.simulation import simulate as ref_simulate  # noqa: E402
from hybridq.extras.gate import Gate as ExtraGate  # noqa: E402
from hybridq.extras.random import get_rqc  # noqa: E402
from hybridq.gate import Gate, Measure, Projection  # noqa: E402

assert refsim._log2_pack_size == 3, 'reference core not found: set LD_LIBRARY_PATH=oracle/_ref'
import device_double  # noqa: E402

device_double.install(setattr)
from hybridq_amd.simulation import simulate  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
    return np.abs(a - b).max() / np.abs(b).max()


def both(circuit, tol=1e-11, **kw):
    kw.setdefault('complex_type', 'complex128')
    kw.setdefault('optimize', 'evolution-hybridq')
    r = ref_simulate(Circuit(circuit), **kw)
    o = simulate(list(circuit), **kw)
    assert o.shape == r.shape and o.dtype == r.dtype, (o.shape, r.shape, o.dtype, r.dtype)
    assert rel(o, r) < tol, rel(o, r)
    return r


rng = np.random.default_rng(7)
np.random.seed(7)
n = 12
base = list(get_rqc(n, 70, use_random_indexes=False))
init = ''.join(rng.choice(list('01+-'), size=n))

# 1. named gates with parameters, powers, conj / T, MATRIX gates
named = [Gate('H', qubits=[q]) for q in range(n)]
named += [Gate('CZ', qubits=[0, 5]), Gate('RZ', qubits=[3], params=[0.37])**1.5, Gate('ISWAP', qubits=[2, 9]).conj(),
          Gate('U3', qubits=[7], params=[0.1, 0.2, 0.3]).T(), Gate('FSIM', qubits=[4, 11], params=[0.4, 0.9]),
          Gate('CPHASE', qubits=[1, 6], params=[1.1])**-1, Gate('SQRT_X', qubits=[8]), Gate('I', qubits=[10]),
          Gate('MATRIX', qubits=[10, 2], U=rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4)))]
both(named + base, initial_state=init)
both(named + base, initial_state=init, compress=0, simplify=False, remove_id_gates=False)
both(named + base, initial_state=init, complex_type='complex64', tol=2e-5)
print('named / matrix / powers: ok')

# 2. string and tuple labels
relabel = {q: (f'q{q}' if q % 2 else (1, q)) for q in range(n)}
lab = [Gate('MATRIX', qubits=[relabel[q] for q in g.qubits], U=g.matrix()) for g in base]
both(lab, initial_state=init)
print('string / tuple labels: ok')

# 3. TupleGates, StochasticGate under a seed
# (one level: the reference's utils.flatten does not descend into a TupleGate inside a TupleGate and then fails on it)
tup = [Gate('TUPLE', gates=base[i:i + 5]) for i in range(0, 40, 5)] + [Gate('TUPLE', gates=base[40:60])] + base[60:]
r = both(tup, initial_state=init, compress=0, simplify=False)
assert rel(r, both(base, initial_state=init, compress=0, simplify=False)) < 1e-12
cand = list(get_rqc(n, 6, use_random_indexes=False))
p = rng.random(len(cand))
stoc = Gate('STOC', gates=cand, p=p / p.sum())
seen = set()
for seed in (1, 2, 3, 4):
    seen.add(both(base[:30] + [stoc] + base[30:], initial_state=init, compress=0, simplify=False, allow_sampling=True,
                  sampling_seed=seed).tobytes())
assert len(seen) > 1
print('tuple / stochastic gates: ok')

# 4. the reference's own FunctionalGates: Projection (host numpy code applied to this driver's state), MessageGate
uni = list(get_rqc(n, 60, use_random_indexes=False, use_unitary_only=True))
both(uni[:30] + [Projection(state='10', qubits=[4, 9])] + uni[30:], initial_state=init, compress=4, simplify=False)
buf_r, buf_o = io.StringIO(), io.StringIO()
r = ref_simulate(Circuit(x for i, g in enumerate(uni) for x in (g, ExtraGate('MESSAGE', qubits=tuple(), message=f'{i}', file=buf_r))),
                 initial_state=init, optimize='evolution-hybridq', complex_type='complex128')
o = simulate([x for i, g in enumerate(uni) for x in (g, ExtraGate('MESSAGE', qubits=tuple(), message=f'{i}', file=buf_o))],
             initial_state=init, optimize='evolution-hybridq', complex_type='complex128')
assert rel(o, r) < 1e-11 and sorted(buf_r.getvalue().split()) == sorted(buf_o.getvalue().split()) and len(buf_o.getvalue().split()) == len(uni)
# Measure: numpy's global generator seeded right before the call on both sides; the reference's simulate() draws other random
# numbers before it reaches the gate, so outcomes are compared through the state they leave behind, outcome by outcome
m_ref = Measure(qubits=[2, 7])
np.random.seed(11)
r = ref_simulate(Circuit(uni + [m_ref]), initial_state=init, optimize='evolution-hybridq', complex_type='complex128', simplify=False)
for s in range(40):  # some seed reproduces the reference's outcome; the collapsed state must then be the reference's
    np.random.seed(s)
    o = simulate(uni + [Measure(qubits=[2, 7])], initial_state=init, optimize='evolution-hybridq', complex_type='complex128', simplify=False)
    if rel(o, r) < 1e-11:
        break
else:
    raise AssertionError('no outcome of the measurement reproduces the reference state')
print('reference FunctionalGates (Projection, Message, Measure): ok')

# 4b. tests.py:2037-2110 (test_simulation_2__fn): half of the gates turned into the reference's FunctionalGates that call the
# reference's own dot() on the split array they are handed -- here that array comes from this driver
from hybridq.utils.dot import dot as ref_dot  # noqa: E402


def as_fn(gate):
    qubits, U = gate.qubits, gate.matrix()

    def f(self, psi, order):
        if not isinstance(psi, np.ndarray):
            raise ValueError("Only 'numpy.ndarray' are supported.")
        axes = [next(i for i, y in enumerate(order) if y == x) for x in qubits]
        return ref_dot(a=U, b=psi, axes_b=axes, b_as_complex_array=psi.ndim > len(order), inplace=True), order
    return Gate('fn', qubits=qubits, f=f)


mixed = [as_fn(g) if rng.random() < 0.5 else g for g in base]
r = ref_simulate(Circuit(base), initial_state=init, optimize='evolution-hybridq', complex_type='complex64')
for opt in ('evolution-hybridq', 'evolution-einsum', 'evolution'):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        o = simulate(mixed, initial_state=init, optimize=opt, complex_type='complex64')
    assert rel(o, r) < 2e-5, (opt, rel(o, r))
print('reference FunctionalGates calling the reference dot(): ok')

# 5. the reference's noisy SuperCircuit (KrausSuperGate objects) through hybridq_amd.dm.simulate
import hybridq.dm.circuit.simulation as ref_dm  # noqa: E402
from hybridq.noise.utils import add_depolarizing_noise  # noqa: E402
from hybridq_amd import dm  # noqa: E402
nq = 6
cq = get_rqc(nq, 24, use_random_indexes=False)
while len(cq.all_qubits()) != nq:
    cq = get_rqc(nq, 24, use_random_indexes=False)
noisy = add_depolarizing_noise(cq, probs=(0.02, 0.05))
init_q = ''.join(rng.choice(list('01+-'), size=nq))
r = ref_dm.simulate(noisy, initial_state=init_q, optimize='evolution-hybridq', complex_type='complex128', verbose=False)
o = dm.simulate(list(noisy), initial_state=init_q, complex_type='complex128', optimize='evolution-hybridq')
assert np.asarray(o).shape == np.asarray(r).shape and rel(o, r) < 1e-11, rel(o, r)
print('reference SuperCircuit through dm.simulate: ok')
print('ALL OK')
